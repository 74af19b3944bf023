#!/bin/bash
# round 4: the team-of-workgroups prototype of the row chains (tools/exp/row_team.hip), built beside the product library
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o tools/exp/librow_team.so tools/exp/row_team.hip \
    -L sparsebev_amd/csrc -l:libsbev_hip.so -Wl,-rpath,$R/sparsebev_amd/csrc
timeout 120 python tools/exp/r4_team_proto.py 900
