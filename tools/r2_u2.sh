python tools/u_debug.py 1 128 1024 11111111 1 8 2>&1 | grep -v amdgpu | tail -12
python tools/u_debug.py 1 128 1024 00000000000000000000 1 20 2>&1 | grep -v amdgpu | tail -24
python tools/u_debug.py 1 128 1024 11100000000111000 1 17 2>&1 | grep -v amdgpu | tail -24
