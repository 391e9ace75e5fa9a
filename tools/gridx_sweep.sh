# effect of the launched grid (multiples of the resident set) on the overlap-save, real-transform,
# filter and any-length kernels
for g in 1 4 16 64 128 100000; do
  echo "== DSPMI_GRIDX=$g"
  DSPMI_GRIDX=$g timeout -k 10 100 python -u tools/ols_one.py 127 1024 512 1048576 || exit 1
  DSPMI_GRIDX=$g timeout -k 10 100 python -u tools/real_probe.py | grep -E "n= *(1024|2048|4096)" || exit 1
  DSPMI_GRIDX=$g timeout -k 10 100 python -u tools/bluestein_probe.py | grep -E "n= *480" | cut -c1-110 || exit 1
done
