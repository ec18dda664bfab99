#!/bin/bash
# what a gpurun box exposes about clocks / power / partition modes (round 4)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/r04_probe
mkdir -p $O
(rocm-smi --showallinfo --json > $O/rocm_smi_all.json) 2> $O/rocm_smi_all.err
rocm-smi -c -P -M -p --showcomputepartition --showmemorypartition > $O/rocm_smi.txt 2>&1
rocm-smi --showmetrics > $O/rocm_smi_metrics.txt 2>&1
amd-smi static --json > $O/amd_smi_static.json 2> $O/amd_smi_static.err
amd-smi metric --json > $O/amd_smi_metric.json 2> $O/amd_smi_metric.err
for c in /sys/class/drm/card*/device; do
  echo "== $c"
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk current_compute_partition current_memory_partition power_dpm_force_performance_level; do
    echo "-- $f"; cat $c/$f 2>&1
  done
  ls $c/hwmon/*/ 2>&1 | head -40
  for f in $c/hwmon/*/power1_cap $c/hwmon/*/power1_average $c/hwmon/*/power1_input $c/hwmon/*/freq1_input $c/hwmon/*/freq2_input; do echo "-- $f"; cat $f 2>&1; done
done > $O/sysfs.txt 2>&1
nproc > $O/host.txt; lscpu | head -30 >> $O/host.txt; free -g >> $O/host.txt
cd $R
python bench.py > $O/bench0.json 2> $O/bench0.err
tail -c 600 $O/bench0.json
