export TMPDIR=/tmp; cd /tmp
for k in 6; do
rm -rf /tmp/rp_s$k
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/rp_s$k -- python /root/repo/scripts/run_variant.py /root/repo/pindel_amd/libpindel_pg_stop$k.so 1000000 > /tmp/rp_s$k.log 2>&1
echo "== STOP $k"; python /root/repo/scripts/pmc_brief.py /tmp/rp_s$k 1000000
done
