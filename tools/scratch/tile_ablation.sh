# Ablation builds of the tile kernels (tile_layer.o only; everything else as built): tools/scratch/libs/{noload,nomfma}.so
set -e
cd /root/repo/trafficbotsv1.5_amd/csrc
mkdir -p ../../tools/scratch/libs
for v in NOLOAD NOMFMA; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DTBX_ABL_$v -c tile_layer.hip -o /tmp/tile_layer_$v.o
  objs=$(ls *.o | grep -v "_clk.o" | grep -v "^tile_layer.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/tile_layer_$v.o -o ../../tools/scratch/libs/$(echo $v | tr A-Z a-z).so
done
ls -la ../../tools/scratch/libs
