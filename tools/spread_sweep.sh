for kb in 48 36 24 16; do for wg in 512 768 1024; do
MDX_SPREAD_LDS_KB=$kb MDX_SPREAD_MIN_WG=$wg bash tools/ab_args.sh "kb${kb}_wg${wg}=--no-output --steps 2000"
done; done
