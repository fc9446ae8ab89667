#!/bin/bash
# Round-5 GPU call: store policy / pieces in flight of the LayerNorm-fold producer epilogue, build-time variants of one source
# (esm_amd/lib/variants/libesmk_{h16plain,allplain,rd4}.so = -DESMK_LNF_H16_NT=0, + -DESMK_LNF_X_NT=0, -DESMK_LNF_RD=4), same box.
set -u
O=gpurun_out/r5b3
mkdir -p $O
LIB=esm_amd/lib/libesmk.so
cp $LIB /tmp/lib_base.so
line() {
  tag=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/$tag.log 2>&1
  grep '^{' $O/$tag.log > $O/$tag.json
  python - "$tag" "$O/$tag.json" <<'PY'
import sys, json
try:
    r = json.loads(open(sys.argv[2]).read())
    kc = {k: v['ms_per_step'] for k, v in r.get('kernel_classes', {}).items() if v['ms_per_step'] > 0.3}
    print(sys.argv[1], r['value'], r['ms_per_step'], 'ms', r.get('library', {}).get('src_hash'), kc, flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e, flush=True)
PY
}
T0=$(date +%s)
line base_b64
for v in h16plain allplain rd4; do
  cp esm_amd/lib/variants/libesmk_$v.so $LIB
  line ${v}_b64
  line ${v}_b4 --batch 4 --steps 20 --warmup 5
done
cp /tmp/lib_base.so $LIB
line base2_b64
line base_b4 --batch 4 --steps 20 --warmup 5
echo "total $(( $(date +%s) - T0 )) s"
