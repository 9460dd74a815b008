#!/bin/bash
# Diagnostics: the stop-ladder builds (-DPG_STOP=k, bench-only instantiations), 8 compiles at a time.
cd "$(dirname "$0")/.." || exit 1
pts="${*:-1 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30 31 32}"
printf '%s\n' plain $pts | xargs -P 8 -I{} bash -c 'if [ {} = plain ]; then bash scripts/build_variant.sh plain -DPG_ONLY_BENCH; else bash scripts/build_variant.sh stop{} -DPG_ONLY_BENCH -DPG_STOP={}; fi; echo built {} $?'
