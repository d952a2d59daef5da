cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for pf in 0 1; do for parts in 512 768 1024; do
IVH_LNL2_PF=$pf IVH_BWD_PARTS=$parts timeout 200 python tools/bench_decoder_tail.py 2>&1 | grep -v amdgpu.ids
done; done
