set -e
echo == old; ./tools/mid_probe_old 5000 | sed -n 1,6p; ./tools/mid_probe_old 5000 | sed -n 30,34p
echo == new; ./tools/mid_probe 5000 | sed -n 1,6p; ./tools/mid_probe 5000 | sed -n 30,34p
