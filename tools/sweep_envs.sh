for v in e2e indi; do for n in 4096 16384 65536 262144 1048576 4194304; do
  k=1000; if [ $n -ge 1048576 ]; then k=100; fi
  python bench.py --variant $v --envs $n --steps $k --warmup 20 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', $n, 'us/step %.2f'%(d['ms_per_step']*1e3), 'Gsteps/s %.2f'%(d['value']/1e9), 'algGB/s %.0f'%(d['roofline']['bytes_per_launch']/d['ms_per_step']/1e6))"
done; done

