#!/bin/bash
# Diagnostics: pindel_pg -i on a synthetic coordinate-sorted BAM at scale (scripts/synth_bam.cpp), with per-stage timing.
#   scripts/bam_scale.sh [ref_len=100000000] [n_pairs=10000000] [extra pindel_pg flags...]
root=$(cd "$(dirname "$0")/.." && pwd)
ref=${1:-100000000}; pairs=${2:-10000000}; shift 2
mkdir -p /tmp/bamscale
[ -x /tmp/synth_bam ] || g++ -O2 -std=c++17 -pthread "$root/scripts/synth_bam.cpp" -lz -o /tmp/synth_bam || exit 1
/usr/bin/env time -v true 2>/dev/null
s=$(date +%s.%N)
/tmp/synth_bam /tmp/bamscale/s "$ref" "$pairs" 150 7 || exit 1
echo "generated in $(echo "$(date +%s.%N) - $s" | bc 2>/dev/null || python3 -c "import time;print('?')") s"
ls -la /tmp/bamscale/s.bam
s=$(date +%s.%N)
PGH_TIMING=1 "$root/pindel_amd/pindel_pg" -f /tmp/bamscale/s.fa -i /tmp/bamscale/s.cfg -o /tmp/bamscale/out "$@" > /tmp/bamscale/log.txt 2> /tmp/bamscale/err.txt
rc=$?
python3 -c "import time,sys; print('pindel_pg -i: rc', $rc, 'wall', round(time.time() - $s, 2), 's')"
grep -E "^pindel_pg:" /tmp/bamscale/log.txt | tail -6
grep "pgh timing" /tmp/bamscale/err.txt
ls -la /tmp/bamscale/out_* | awk '{print $5, $9}'
