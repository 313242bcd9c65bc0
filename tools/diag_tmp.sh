for env in "" "NW_BC_PRE=1" "NW_BC_PRE=1 NW_BC_MOMENTS=1"; do
for hw in 14 7; do
echo "== hw $hw env [$env]"
env $env ./tools/bench_conv_plain 42 128 $hw $hw 32 3 | tail -1
env $env ./tools/bench_conv 42 128 $hw $hw 32 3 | tail -3
done; done
