set -e
mkdir -p gpurun_out
for s in 64 16; do
  echo "== sub $s" >> gpurun_out/lanes_ab.txt
  for n in 1 2 4 8 16 64 256; do
  PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_sub$s.so timeout -k 10 300 python3 tools/hybrid_batch_rate.py $n 8 2>&1 | grep "^N" >> gpurun_out/lanes_ab.txt
  done
done
cat gpurun_out/lanes_ab.txt
