{
echo "# tools/stress_parity.py (round 6, HEAD of the round): random cameras, GPU visible sets / records / isVisible against the oracle, bit for bit; plain, block-bounds and RG16F contexts"
for a in "--entities 10000000 --views 9 --depth 4096x4096 --depth-kind noise --hiz-share 1 --seed 51" "--entities 10000000 --views 12 --depth 4096x4096 --seed 52" "--entities 2000000 --views 36 --seed 53" "--entities 2000000 --views 36 --depth 1920x1080 --depth-kind noise --hiz-share 2 --seed 54" "--entities 500000 --views 48 --depth 1366x768 --seed 55"; do
  echo "## $a"
  timeout 900 python3 tools/stress_parity.py $a 2>&1 | grep -v amdgpu.ids | tail -4
done
} > gpurun_out/r06_stress_parity.txt 2>&1
tail -30 gpurun_out/r06_stress_parity.txt
