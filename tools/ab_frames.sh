for rep in 1 2 3; do for lib in ${AB_LIBS:-tools/exp_old.so tools/exp_new.so}; do
RF_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --cpu-rays 0 --dropin-steps 0 --render-frames 8 --highres-frames 4 --windows 0 --second-point-rays 0 --train256-steps 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
f=l['fwd_render']; h=l['highres_render']
def g(d,*ks):
    for k in ks:
        d=d.get(k,{}) if isinstance(d,dict) else {}
    return d
print('$lib','step %.4f'%l['ms_per_step'],'fwdpair %.4f'%l['kernels']['render_forward[spec+diffuse,save]']['avg_ms'],'frame init',f['init_field'].get('kernel_ms_per_frame'),f['init_field'].get('ms_per_frame'),'trav',f['traversal'].get('kernel_ms_per_frame'),f['traversal'].get('ms_per_frame'), 'highres', [ (k, v.get('ms_per_frame')) for k,v in h.items() if isinstance(v,dict) and 'ms_per_frame' in v])
"
done; done
