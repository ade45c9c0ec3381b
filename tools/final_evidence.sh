set -x
cd $GRAFT_REPO_ROOT
( time python bench.py > gpurun_out/r05_f_default_stdout.txt 2> gpurun_out/r05_f_default_stderr.txt ) 2>&1 | tail -4
tail -1 gpurun_out/r05_f_default_stdout.txt | cut -c1-300
bash tools/refresh_profiles.sh r05_f > gpurun_out/r05_f_refresh.log 2>&1
tail -30 gpurun_out/r05_f_refresh.log
bash tools/gather_pmc.sh r05_f > gpurun_out/r05_f_gather_pmc.log 2>&1
tail -5 gpurun_out/r05_f_gather_pmc.log
