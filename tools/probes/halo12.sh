cd /root/repo
python tools/shape_profile.py 2>&1 | grep "conv3x3\|total\|reduce" > gpurun_out/halo_def.txt
VD_CONV_HALO=12 python tools/shape_profile.py 2>&1 | grep "conv3x3\|total\|reduce" > gpurun_out/halo_12.txt
python tools/halo_forward.py "-1 12" 2>&1 | grep round
