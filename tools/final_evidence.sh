set -x
cd $GRAFT_REPO_ROOT
( time python bench.py > gpurun_out/r05_h_default_stdout.txt 2> gpurun_out/r05_h_default_stderr.txt ) 2>&1 | tail -4
tail -1 gpurun_out/r05_h_default_stdout.txt | cut -c1-300
bash tools/refresh_profiles.sh r05_h > gpurun_out/r05_h_refresh.log 2>&1
tail -30 gpurun_out/r05_h_refresh.log
bash tools/gather_pmc.sh r05_h > gpurun_out/r05_h_gather_pmc.log 2>&1
tail -5 gpurun_out/r05_h_gather_pmc.log
