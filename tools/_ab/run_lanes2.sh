set -e
mkdir -p gpurun_out
for s in 16 8 4; do
  echo "== sub $s" >> gpurun_out/lanes_ab2.txt
  for n in 1 16 64 4096 65536; do
  PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_sub$s.so timeout -k 10 200 python3 tools/hybrid_batch_rate.py $n 8 2>&1 | grep "^N" >> gpurun_out/lanes_ab2.txt
  done
done
cat gpurun_out/lanes_ab2.txt
