import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print(d['step'], d['ms_per_step'], 'pi', d['pressure_iter_us'], 'V', d['pressure_vel_loop_us'], 'D', d['pressure_drho_loop_us'], 'dens', d['density_coff_us'], 'grid', d['grid_update_us'], 'bl', d['build_lists_us'], 'visc', d['visc_iter_us'])
    elif l.startswith('##') or 'passed' in l or 'failed' in l or 'GPU-minutes' in l: print(l)
