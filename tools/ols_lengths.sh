# which transform length is fastest per tap count (the plan picks the smallest power of two >= 4*nh)
for cfg in "255 1024" "255 2048" "511 2048" "511 4096" "1023 4096" "1023 8192" "2047 8192" "2047 16384" "4095 16384" "4095 8192"; do
  timeout -k 10 100 python -u tools/ols_one.py $cfg 512 1048576 || exit 1
done
