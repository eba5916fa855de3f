# round 4: second set of knock-out builds of scan_plain_wave_kernel (WRONG results on purpose, micro only; the switch
# lived in plain_scan.hip only for this measurement): what does a wave's time consist of?
# 1 = no epilogue arithmetic (clamp / pack / minima / swaps), 2 = no flush (LDS tile -> global), 4 = no staging stores,
# 8 = one table row instead of 26 per unit, 16 = chain of 2 MFMAs (and 2 one-hot reads) instead of 26,
# 32 = one-hot operand from registers (address arithmetic kept, no LDS read), 64 = two VALU adds instead of each MFMA; sums combine
O=gpurun_out/r04/plain_knock2b.txt; mkdir -p gpurun_out/r04; : > $O
for k in 0 32 64 96 0; do
  echo "== knock $k" >> $O
  timeout -k 5 60 scripts/micro/bin/mfma_scan_kn$k 1087 69 10000 9 512 0 12 1 2>&1 | grep "scan_plain_kernel:" | cut -c1-60 >> $O
done
cat $O
