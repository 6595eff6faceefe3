for g in 2048 1792 1536 1280 1250 1024 840 834; do echo "trace grid $g: $(SD_TRACE_GRID=$g python bench.py --steps 6 --no-cpu-baseline --timed-only --pipe-mode 0 2>/dev/null | python3 -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step' % j['ms_per_step'], j['kernel_ms_per_step'])")"; done
for g in 512 448 417 420 384 334 340; do echo "fill grid $g: $(SD_FILL_GRID=$g python bench.py --steps 6 --no-cpu-baseline --timed-only --pipe-mode 0 2>/dev/null | python3 -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step' % j['ms_per_step'], j['kernel_ms_per_step'])")"; done
