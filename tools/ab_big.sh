# long transforms: base library against the current one (c2c by length, overlap-save with long filters)
for v in base new base new; do
  if [ $v = base ]; then export DSPMI_LIB=$PWD/tools/ab/libdspmi_base.so; else unset DSPMI_LIB; fi
  echo "== $v"
  timeout -k 10 100 python -u tools/fft_sizes.py | grep -E "n= *(4096|8192|16384)" || exit 1
  timeout -k 10 100 python -u tools/ols_one.py 2047 8192 64 4194304 || exit 1
  timeout -k 10 100 python -u tools/ols_one.py 8191 16384 1 67108864 || exit 1
done
