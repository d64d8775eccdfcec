mkdir -p gpurun_out/r02i
python tools/rate_other_fs.py > gpurun_out/r02i/rate_base.log 2>&1
GPSACQ_WIDE3=1 python tools/rate_other_fs.py > gpurun_out/r02i/rate_wide3.log 2>&1
cat gpurun_out/r02i/rate_base.log gpurun_out/r02i/rate_wide3.log
