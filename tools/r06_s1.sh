cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_chainpmc; mkdir -p $O
for v in base ldsw base ldsw; do
  rm -rf /tmp/st_$v; ISDF_HIP_LIB=$R/variants/lib_$v.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$v -- python $R/tools/train_only.py 200 > /dev/null 2>&1
  f=$(find /tmp/st_$v -name '*kernel_stats.csv' | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'chain_kernel' in r['Name'] or 'dw_kernel' in r['Name']: print('$v', r['Name'][:40], r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))"
done > $O/whatif_lds_write.txt; cat $O/whatif_lds_write.txt
