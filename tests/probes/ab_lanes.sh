for cfg in "384 2 10" "384 2 10 0" "384 3 10" "512 2 10" "384 4 10"; do
  ENLSIP_GN_PIPELINE=0 python tests/probes/two_lane_probe.py $cfg 2>/dev/null
done
python tests/probes/two_lane_probe.py 384 1 10 2>/dev/null
python tests/probes/two_lane_probe.py 768 2 10 2>/dev/null
