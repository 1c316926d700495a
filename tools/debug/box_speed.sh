# which box is this?  clocks / power cap as rocm-smi reports them, then the headline step alone
rocm-smi --showpower --showmaxpower --showclocks --showperflevel --showtemp 2>&1 | grep -v "^$" | head -40
python bench.py --no-extra --no-cpu --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms_per_step', d['ms_per_step'], d['msm_phase_ms'])"
rocm-smi --showpower --showclocks 2>&1 | grep -iE "sclk|power" | head -6
