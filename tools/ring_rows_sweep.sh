for cfg in "10 16 16 8" "9 16 16 8" "8 20 20 10" "8 24 24 8" "6 24 24 12" "6 32 32 16" "7 24 24 8" "12 16 16 8" "11 16 16 8" "9 20 20 10" "7 20 20 10" "10 12 12 6"; do
  set -- $cfg
  echo "== L $1 nq $2 rows $3 vs $4 vs default"
  timeout -k 10 300 python tools/ab_libs.py libshpair.so libshpair.so libshpair.so --ring-rows $3 $4 0 --split 0 0 0 --lmax $1 --nq $2 --rounds 4 2>&1 | grep median
done
