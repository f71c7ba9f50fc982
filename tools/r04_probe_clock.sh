ls /sys/class/drm/ 2>&1 | head; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done 2>&1 | head -20
ls /sys/class/drm/card*/device/hwmon/*/freq1_input 2>&1 | head -3; cat /sys/class/drm/card*/device/hwmon/*/freq1_input 2>&1 | head -3
time rocm-smi --showclocks 2>&1 | head -20
python -c "import amdsmi; print('amdsmi ok')" 2>&1 | tail -1
which amd-smi rocm-smi
time rocm-smi --showclocks --json 2>&1 | head -5
