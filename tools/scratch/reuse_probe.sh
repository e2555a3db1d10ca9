root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out; rm -f gpurun_out/reuse.txt
for cfg in "TBX_FUSED_TAIL=1" "TBX_FUSED_TAIL=0" "TBX_FUSED_TAIL=1 TBX_NAVI_RIDER=0"; do
  echo "[$cfg]" >> gpurun_out/reuse.txt
  env $cfg python bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --new-scenes 4 2>&1 | grep '"value"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scene_reuse']['new_scene_ms_all'], d['value'])" >> gpurun_out/reuse.txt
done
cat gpurun_out/reuse.txt
