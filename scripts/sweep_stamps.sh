# One parametrised sweep over a knob of the DEVELOPMENT build (libfoho_hip_stamps.so: `make -C followmyhold_amd/csrc STAMPS=1`,
# the only build that reads FOHO_DEBUG_* from the environment), on the GPU box:
#     bash scripts/sweep_stamps.sh FOHO_DEBUG_RFH "1 2 4 8" --crop hoi --steps 300
#     bash scripts/sweep_stamps.sh FOHO_DEBUG_GTILES "128 256 512 1024" --crop hoi --images 32 --streams 4 --steps 100
# Knobs: FOHO_DEBUG_RFH / _RFO (hand / object faces per raster workgroup), _GTILES / _GFRAC (k_pix_bwd workgroups), _LEAN
# ("h,o,i" threads kept per raster / inside workgroup), _IFH / _IFO (inside-test faces per workgroup), _SKIP_ROLES (ablation
# bit mask), _RWIN (k_resolve tile window).  Replaces the per-experiment scripts of round 3 (their results: NOTEBOOK.md).
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; shift 2
make -C $R/followmyhold_amd/csrc STAMPS=1 -s 2>&1 | grep -E "error"
for v in $VALS; do
    printf "%s=%s: " $VAR $v
    env FOHO_HIP_SO=$R/followmyhold_amd/libfoho_hip_stamps.so $VAR=$v timeout 300 python $R/scripts/run_steps.py "$@" 2>&1 | grep "steps/s"
done
