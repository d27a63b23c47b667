# ring-group size of the two-wave kernels at the large configurations (round-4 kernels): tools/l12_sweep.sh
for cfg in "12 32" "11 32" "10 32" "12 24"; do
  set -- $cfg
  echo "== L $1 nq $2: two waves per pair, rows 6 8 10 12 16 default"
  timeout -k 10 500 python tools/ab_libs.py libshpair.so libshpair.so libshpair.so libshpair.so libshpair.so libshpair.so --split 1 1 1 1 1 1 --ring-rows 6 8 10 12 16 0 --lmax $1 --nq $2 --rounds 3 2>&1 | grep median
done
for cfg in "12 32" "12 16" "10 24" "9 32"; do
  set -- $cfg
  echo "== L $1 nq $2: split 0 vs 1 (default rows)"
  timeout -k 10 500 python tools/ab_libs.py libshpair.so libshpair.so --split 0 1 --lmax $1 --nq $2 --rounds 3 2>&1 | grep median
done
