# A/B of several builds of libshpair over a matrix of (L, n_q): bash tools/ab_matrix.sh lib1.so lib2.so ...
set -e
LIBS="$*"
for cfg in "6 16 5" "4 10 5" "5 12 5" "9 12 5" "8 16 5" "7 16 5" "3 12 5" "10 16 3" "12 32 3" "12 16 3"; do
  set -- $cfg
  echo "== L $1 nq $2"
  timeout -k 10 400 python tools/ab_libs.py $LIBS --lmax $1 --nq $2 --rounds $3 2>&1 | grep median
done
