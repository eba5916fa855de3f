# round 4: the plain-scan micro benchmark over variants; usage: scripts/r04_micro.sh OUT "cap K variant" ...
# (GloVe-like shape: 1087 lists x ~69 chunks, 10 000 queries x 9 probes; then lists shared by few queries)
O=gpurun_out/r04/$1; shift
mkdir -p gpurun_out/r04; : > $O
B=${MICRO_BIN:-scripts/micro/bin/mfma_scan}
for v in "$@"; do
  set -- $v
  echo "== shape $1 cap $2 K $3 variant $4" >> $O
  if [ "$1" = glove ]; then A="1087 69 10000 9 512"; elif [ "$1" = sparse ]; then A="10000 69 10000 9 512"; else A="$1"; fi
  timeout -k 5 60 $B $A $2 $3 $4 >> $O 2>&1 || echo FAILED >> $O
done
cat $O
