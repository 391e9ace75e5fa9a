#!/bin/bash
# Long transforms under the experiment switches: work-frame group size.
export DSPMI_FFT_BUNDLE=8
for mb in 128 256 512 1024; do
    echo "== work $mb MiB"
    DSPMI_FFT_WORK_MB=$mb timeout -k 10 120 python tools/fft_sizes.py 65536 262144 1048576 || exit 1
done
