for rep in 1 2; do
for cfg in "1 40" "1 8" "4 8" "8 8" "16 8" "0 40"; do
  set -- $cfg
  echo -n "XPF_D=$1 MIN_NK=$2  "
  ESMK_XPF_D=$1 ESMK_XPF_MIN_NK=$2 python tools/microbench.py --only gemm --iters 20 2>/dev/null | grep "resid" | grep "persistent " | awk '{printf "%s %s us | ", $2, $(NF-3)}'
  echo
done; done
