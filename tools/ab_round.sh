for cfg in "6 16 8" "4 10 6" "5 12 5" "3 12 5" "7 16 5" "8 16 5" "9 12 4" "9 16 4" "10 16 4" "12 16 3" "12 32 3" "6 24 4" "6 32 3" "8 24 3"; do
  set -- $cfg
  echo "== L $1 nq $2"
  SHPAIR_AB_OLD_LIB=1 timeout -k 10 400 python tools/ab_libs.py libshpair_6774871.so libshpair.so --lmax $1 --nq $2 --rounds $3 2>&1 | grep median
done
