#!/bin/bash
# time the attention micro-benchmark (B=64, H=20, T=1024) for a list of ESMK_ATTN variants, interleaved twice
for rep in 1 2; do for v in "$@"; do echo -n "ESMK_ATTN=$v  "; ESMK_ATTN=$v python tools/microbench.py --only attn --iters 30 2>/dev/null | tail -1; done; done
