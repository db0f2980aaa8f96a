# Same box, two libraries: bench.py's vae_transformer / pipeline_iteration records on followmyhold_amd/libfoho_hip_old.so (a build of an
# earlier commit, put there by hand) and on the current build.  bash scripts/dev/ab_pipe.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
pick='import json,sys
for l in sys.stdin:
    if not l.startswith("{"): continue
    d=json.loads(l)
    if "fwd_ms" in d: print("   vae_transformer fwd %.2f bwd %.2f ms, four images fwd+bwd %.2f" % (d["fwd_ms"], d["bwd_ms"], d["b4_fwd_bwd_ms"]))
    if "hip_transformer" in d: print("   pipeline_iteration %.2f ms (backward %.2f), four images %.2f, no-gradient decode %.2f" % (d["hip_transformer"]["iteration_ms"], d["hip_transformer"]["backward_ms"], d["batch_of_4"]["iteration_ms"], d.get("step_decode_nograd_ms", float("nan"))))'
run() { python $R/scripts/dev/pipe_iter.py 2>/dev/null | python -c "$pick"; }
echo "== NEW"; run
cp $R/followmyhold_amd/libfoho_hip.so /tmp/new.so; cp $R/followmyhold_amd/libfoho_hip_old.so $R/followmyhold_amd/libfoho_hip.so
echo "== OLD"; run
cp /tmp/new.so $R/followmyhold_amd/libfoho_hip.so
echo "== NEW again"; run
