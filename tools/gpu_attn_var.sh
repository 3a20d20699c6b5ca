#!/bin/bash
export TMPDIR=/tmp
for cfg in "1 2 1" "2 2 1" "1 1 1" "1 2 2"; do set -- $cfg
  echo "NS_FWD=$1 NS=$2 NK=$3: $(CE_ATTN_NS_FWD=$1 CE_ATTN_NS=$2 CE_ATTN_NK=$3 python tools/diag/attn_long_time.py)"
done
