import json,sys
for f in sys.argv[1:]:
    p=json.load(open(f))
    print(f, 'total', round(sum(o['mean_us'] for o in p)))
    for i,o in enumerate(p):
        if (o['name'].endswith('.conv') and 'output_blocks' in o['name']) or o['kind']=='stats_fold':
            print('  ', o['op'], o['name'], o.get('shape'), round(o['mean_us'],1), 'next:', p[i+1]['name'] if i+1<len(p) else '', round(p[i+1]['mean_us'],1) if i+1<len(p) else '')
