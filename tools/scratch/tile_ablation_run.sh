root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root; mkdir -p gpurun_out; rm -f gpurun_out/tile_abl.txt
for v in base noload nomfma; do
  lib=$root/trafficbotsv1.5_amd/csrc/libtbx_hip.so; [ $v != base ] && lib=$root/tools/scratch/libs/$v.so
  TBX_HIP_LIB=$lib python bench.py --agents 128 --rollouts 32 --steps 40 --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --new-scenes 0 2>&1 | tail -1 > gpurun_out/abl_$v.json
  python - <<PY >> gpurun_out/tile_abl.txt
import json
d=json.loads(open("gpurun_out/abl_$v.json").read())
print("$v", round(d["value"]), [ (k["class"], k.get("kernel","")[:24], k["launches_per_step"], round(k["avg_launch_us"],1)) for k in d["kernels"] if k["class"] in ("tile",)])
PY
done
cat gpurun_out/tile_abl.txt
