cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "0 1 256" "1 1 256" "0 0 256" "1 0 256" "0 1 128"; do
  tag=$(echo $v | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace -d $R/gpurun_out/pgv_$tag -o pg -- python3 $R/tools/pg_bench.py 100 $v > $R/gpurun_out/pgv_$tag.log 2>&1 || exit 1
done
