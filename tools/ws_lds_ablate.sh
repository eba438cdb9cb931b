# Is conv_ws_kernel's K loop bound by its LDS fragment reads?  Builds a second library whose conv_ws_kernel reads ONE weight fragment per k-step instead of nine
# (-DVMG_WS_ABL_ONE_FRAG: 3 KiB instead of 11 per wave and k-step, same 18 MFMAs) and times both on the dominant shape.   bash tools/ws_lds_ablate.sh   (here: builds; GPU box: runs)
set -e
cd "$(dirname "$0")/.."
if [ ! -f vmg_amd/libvmg_hip_abl.so ] || [ vmg_amd/csrc/conv_igemm.hip -nt vmg_amd/libvmg_hip_abl.so ]; then
  python -m vmg_amd.build > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DVMG_WS_ABL_ONE_FRAG -c vmg_amd/csrc/conv_igemm.hip -o /tmp/conv_igemm_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vmg_amd/libvmg_hip_abl.so /tmp/conv_igemm_abl.o $(ls vmg_amd/build/*.o | grep -v conv_igemm)
fi
if python -c "import torch, sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
  echo "shipped library:"; python tools/ws_lds_ablate.py
  echo "one weight fragment per k-step:"; VMG_HIP_LIB=vmg_amd/libvmg_hip_abl.so python tools/ws_lds_ablate.py
fi
