lib=$(readlink -f "$1"); reads=2000000
root=/root/repo
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/rp_if
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INSTS_BRANCH \
    --output-format csv -d /tmp/rp_if -- python "$root/scripts/run_variant.py" "$lib" "$reads" 2>/dev/null | grep "kernel ms"
python "$root/scripts/pmc_brief.py" /tmp/rp_if "$reads"
