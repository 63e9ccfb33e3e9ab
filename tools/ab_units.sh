# units-per-product probes of the f16f8 conv (tools/probe_units.py), one process per library: tools/ab_units.sh
mkdir -p gpurun_out
for lib in libwsu.so libwsu_probe3.so libwsu_probe1.so libwsu_probe2.so libwsu_probe4.so libwsu.so; do
  timeout -k 10 200 python tools/probe_units.py $lib 2>&1 | grep "us " || exit 1
done 2>&1 | tee gpurun_out/ab_units.log
