O=gpurun_out/r04m_chain_floor.txt
python tools/_probe_chain.py 2>&1 | grep -v amdgpu.ids > $O
WDG_LIB=$PWD/gpurun_variants/libwdgan_exp.so python tools/_probe_chain.py patch_dbg=119 2>&1 | grep -v amdgpu.ids >> $O
cat $O
