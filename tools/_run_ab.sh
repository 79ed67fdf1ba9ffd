for h in 0 1; do echo "== head $h"; python tools/stream_probe.py --head $h --n 100 2>&1 | grep -E "^stream:|^    |job:" | cut -c1-200 | head -16; done
