for v in old new old new; do
if [ $v = old ]; then export IMF_LIB=$PWD/imfnet_amd/_abl/old.so; else unset IMF_LIB; fi
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['timing']['ms_per_step_min'], d['roofline']['per_kernel']['k_spconv_g<2, 0>']['avg_launch_us'])"
done
