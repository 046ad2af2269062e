#!/bin/bash
# VGPR / spill summary of the skewed fp16 kernels (resource-usage remarks of hipcc)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function $ESN_EXTRA_FLAGS -Rpass-analysis=kernel-resource-usage -c /root/repo/esn_ofdm_mimo_amd/csrc/esn_recur_mfma_f16.hip -o /tmp/f16.o 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|SGPRs Spill" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - | grep "ELb1EEEv\|error" | sed -e 's/_ZN3esn17recur_mfma_kernelINS_9Traits//' -e 's/EEEvNS_11RecurParamsE//'
