# query-major flat scan at HBM scale: loop forms (build-time switches), 1 GiB of codes, nq = 1
R=$PWD
for v in "" "-DTK_FLAT_PREFETCH=1" "-DTK_FLAT_UNROLL=4" "-DTK_FLAT_PREFETCH=1 -DTK_FLAT_UNROLL=4"; do
  (cd tinyknn_amd/csrc && rm -f adc_scan.o && make -s FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result $v" 2>/dev/null)
  echo "variant [$v]"
  python bench_scan.py --log2n 26 --M 32 --nq 1 --reps 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   ', {k: (round(v,3) if isinstance(v,float) else v) for k,v in j.items() if k in ('nq','ms','algorithmic_GBps','kernel')})"
done
(cd tinyknn_amd/csrc && rm -f adc_scan.o && make -s 2>/dev/null)
