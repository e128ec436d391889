mkdir -p gpurun_out/r6t
export N2M_TV_FWD=1
bash tools/env_sweep.sh N2M_FWD_XCD_GROUP "4 2 1 0" r6t/xg 2>&1 | tee gpurun_out/r6t/xg_sweep.txt
unset N2M_TV_FWD
bash tools/env_ab.sh N2M_TV_FWD 0 1 r6t/garden --recipe garden 2>&1 | tee gpurun_out/r6t/tvfwd_garden_ab.txt
