import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
g=collections.OrderedDict()
for r in rows:
    if 'wgrad' in r['Kernel_Name']:
        k=(r['Kernel_Name'][:44], r['Grid_Size_X'], r['Grid_Size_Y'])
        g.setdefault(k,[]).append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in g.items():
    print(k, len(v), 'min %.1f median %.1f' % (min(v)/1e3, sorted(v)[len(v)//2]/1e3))
