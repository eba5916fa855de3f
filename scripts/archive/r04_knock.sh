# round 4: knock-out builds of the wave-per-unit plain kernel (WRONG results on purpose): what bounds it?
# 1 = no stores, 2 = no one-hot LDS reads, 4 = no MFMA, 8 = no minimum-byte stores (sums of those)
O=gpurun_out/r04/$1; shift
mkdir -p gpurun_out/r04; : > $O
for k in ${KNOCKS:-"" _k1 _k8 _k2 _k4 _k3 _k7}; do
  echo "== build mfma_scan$k" >> $O
  timeout -k 5 60 scripts/micro/bin/mfma_scan$k 1087 69 10000 9 512 0 ${1:-12} 1 2>&1 | grep "scan_plain_kernel:" >> $O
done
cat $O
