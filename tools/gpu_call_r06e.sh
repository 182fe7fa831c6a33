#!/bin/bash
# Round 6, probe: the TCP solver's world alone under LDS placements (tools/solver_world_placement_probe.py; the -DRB_LDS_ARENA build under ab_libs/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_arena.so
CONS=cand,con,conj,conidx,row,dofcon,conf,conloc
KIN=xpos,xquat,xipos,xiquat,xanchor,xaxis,gpos,gquat,spos,rootcom,cinert,crb,cdof,cdofdot,cvel,cacc,cfrc,msp,dofcon_adr,cfrcext
{
for p in none kin; do
  timeout 600 python tools/solver_world_placement_probe.py $p 1024 2304 4096 2>&1 | grep -v amdgpu.ids
done
export SOLVER_MAXCON=8 SOLVER_MAXROW=48
for p in none $CONS $KIN,$CONS; do
  timeout 600 python tools/solver_world_placement_probe.py $p 1024 1536 2048 4096 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/solver_world_placement.txt
