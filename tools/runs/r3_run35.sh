set -e
for p in potf2_probe potf2_probe_qplain potf2_probe potf2_probe_qplain; do echo "== $p"; ./tools/$p | tail -11 | head -3; done
