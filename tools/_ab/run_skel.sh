set -e
mkdir -p gpurun_out
for n in 65536 131072 262144 1048576; do echo "== N $n"; timeout -k 5 120 tools/_ab/skeleton_probe $n | grep -E "^empty|step's skeleton|29 in, 41 out  |3.3 us of arith"; done > gpurun_out/skel_sizes.txt 2>&1
cat gpurun_out/skel_sizes.txt
