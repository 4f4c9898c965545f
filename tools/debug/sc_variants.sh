set -e
cd $GRAFT_REPO_ROOT
for v in "-DPP_X_TAIL_DRAIN" "-DPP_X_NOFUSE" ""; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-variable -Wno-unused-but-set-variable $v -c packppi_amd/csrc/pp_edge.hip -o packppi_amd/csrc/pp_edge.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o packppi_amd/csrc/libpackppi_hip.so packppi_amd/csrc/pp_api.o packppi_amd/csrc/pp_prepare.o packppi_amd/csrc/pp_node.o packppi_amd/csrc/pp_edge.o packppi_amd/csrc/pp_clash.o
  echo "== variant [$v]"
  python tools/debug/score_check3.py 400 739 1024 2>&1 | grep "L="
  for i in 1 2 3; do python tools/debug/edge_repro.py 1024 2>&1 | grep "^rep" | cut -c1-100 | grep -v "hE: 0 \[\] | differing S: 0" | wc -l; done | tr "\n" " "; echo
done
