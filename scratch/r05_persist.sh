#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/lab
hipcc --offload-arch=gfx950 -O3 persist_chain.hip -o /tmp/persist_chain 2>&1 | grep -i error
timeout 120 /tmp/persist_chain 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r05_persist_chain.log
