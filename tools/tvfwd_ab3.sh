mkdir -p gpurun_out/r6u
export N2M_TV_FWD=1
bash tools/env_ab.sh N2M_PM_SPLIT 0 1 r6u/split 2>&1 | tee gpurun_out/r6u/tvfwd_split_ab.txt
export N2M_PM_SPLIT=1
bash tools/env_sweep.sh N2M_PM_FINE_GROUPS "96 64 128" r6u/groups 2>&1 | tee gpurun_out/r6u/tvfwd_split_groups.txt
