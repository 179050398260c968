# phase clocks of k_tr_entries, K9 and k_sn_panel (profiling build made ON THE BOX with make K9_TIMING=1; the shipped library of the snapshot is rebuilt there, nothing is committed):
#   gpurun --timeout 1500 -- bash tools/measure_r04_phase_clocks.sh   ->  profiles/r04_{tr,k9,sn_panel}_phase_times.txt
ulimit -c 0
D=gpurun_out/${1:-r04t}; mkdir -p $D
touch linearsfm_amd/csrc/lsfm_pcg.hip linearsfm_amd/csrc/lsfm_schur_panel.hip linearsfm_amd/csrc/lsfm_transform.hip
make -s -C linearsfm_amd/csrc K9_TIMING=1 -j8 > $D/build.log 2>&1
timeout 600 python tools/tr_phase_times.py > $D/tr_phase.txt 2>&1
timeout 600 python tools/k9_phase_times.py > $D/k9_phase.txt 2>&1
timeout 600 python tools/sn_phase_times.py nc3500 > $D/sn_phase_nc3500.txt 2>&1
cat $D/tr_phase.txt $D/k9_phase.txt $D/sn_phase_nc3500.txt
