#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c10; rm -rf $O; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/scorer_lab.hip -Imodels_amd/csrc -o /tmp/scorer_lab
echo "grid data" | tee $O/lab.txt; /tmp/scorer_lab | tail -2 | tee -a $O/lab.txt
echo "dense data" | tee -a $O/lab.txt; SCORER_LAB_DATA=dense /tmp/scorer_lab | tail -2 | tee -a $O/lab.txt
echo "grid data" | tee -a $O/lab.txt; /tmp/scorer_lab | tail -2 | tee -a $O/lab.txt
