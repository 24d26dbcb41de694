set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c2 t1 c3 c4 c5 echo; do
  timeout 500 bash tools/pmc_pass.sh $w r06z_$w > /dev/null 2>&1
  head -8 gpurun_out/r06z_${w}_stats.txt
done
