for fl in "" "sum" "clone" "slice" "kcopy" "churn" "sum clone slice churn"; do python tools/scratch/graph_sum2.py 3000 $fl 2>&1 | tail -1; done
echo "--- DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
for fl in "clone" "slice" "sum clone slice churn"; do DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/scratch/graph_sum2.py 3000 $fl 2>&1 | tail -1; done
echo "--- AMD_SERIALIZE_COPY=3"
for fl in "clone" "sum clone slice churn"; do AMD_SERIALIZE_COPY=3 python tools/scratch/graph_sum2.py 3000 $fl 2>&1 | tail -1; done
echo "--- small graph (200 its)"
for fl in "clone" "sum clone slice churn"; do python tools/scratch/graph_sum2.py 200 $fl 2>&1 | tail -1; done
