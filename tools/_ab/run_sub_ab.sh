set -e
mkdir -p gpurun_out
for s in 64 32 16; do
  echo "== sub $s" >> gpurun_out/sub_ab.txt
  PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_sub$s.so timeout -k 10 300 python3 tools/hybrid_batch_rate.py 65536 6 >> gpurun_out/sub_ab.txt 2>&1
  PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_sub$s.so timeout -k 10 300 python3 tools/hybrid_batch_rate.py 4096 6 >> gpurun_out/sub_ab.txt 2>&1
done
echo "== a4 numbers, sub 16" >> gpurun_out/sub_ab.txt
PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_sub16.so timeout -k 10 400 python3 tools/a4_hybrid_numbers.py integrate >> gpurun_out/sub_ab.txt 2>&1
echo "== a4 numbers, sub 64" >> gpurun_out/sub_ab.txt
PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_sub64.so timeout -k 10 400 python3 tools/a4_hybrid_numbers.py integrate >> gpurun_out/sub_ab.txt 2>&1
cat gpurun_out/sub_ab.txt
