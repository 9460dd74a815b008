cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden_pin.py -x -q -m gpu 2>&1 | tail -3
python scripts/bench_brief.py --no-cpu-baseline --no-host-path --reads 4000000 --steps 3
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/rp1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d /tmp/rp1 -- python /root/repo/scripts/run_variant.py /root/repo/pindel_amd/libpindel_pg.so 2000000 > /tmp/rp1.log 2>&1
tail -1 /tmp/rp1.log
python /root/repo/scripts/pmc_brief.py /tmp/rp1 2000000
