cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { rm -rf /tmp/vt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vt -o vt -- python $R/scripts/dev/vae_trace.py 1 12 > /dev/null 2>&1; python $R/scripts/dev/trace_summary.py $(find /tmp/vt -name "*kernel_trace.csv" | head -1) k_vae_rowstat 2>&1 | head -${1:-20}; }
python $R/scripts/dev/vae_trace.py 1 3 > /dev/null 2>&1
echo "== NEW"; run 14
cp $R/followmyhold_amd/libfoho_hip.so /tmp/new.so; cp $R/followmyhold_amd/libfoho_hip_old.so $R/followmyhold_amd/libfoho_hip.so
echo "== OLD"; run 20
cp /tmp/new.so $R/followmyhold_amd/libfoho_hip.so
echo "== NEW again"; run 3
