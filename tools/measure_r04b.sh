# phase clocks of the supernode panel kernel (profiling build made on the box; the shipped library is not touched)
ulimit -c 0
D=gpurun_out/r04b; mkdir -p $D
touch linearsfm_amd/csrc/lsfm_pcg.hip linearsfm_amd/csrc/lsfm_schur_panel.hip linearsfm_amd/csrc/lsfm_transform.hip
make -s -C linearsfm_amd/csrc K9_TIMING=1 -j8 > $D/build.log 2>&1
timeout 600 python tools/sn_phase_times.py nc3500 > $D/sn_phase_nc3500.txt 2>&1
timeout 900 python tools/sn_phase_times.py synth16k > $D/sn_phase_synth16k.txt 2>&1
cat $D/sn_phase_nc3500.txt $D/sn_phase_synth16k.txt
