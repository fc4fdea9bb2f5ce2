import csv,sys
rows=list(csv.DictReader(open('gpurun_out/prof_delta/kt_kernel_trace.csv')))
idx=[i for i,r in enumerate(rows) if 'k_delta_mark' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
tot=0
for r in rows[i0:i0+100]:
    n=r['Kernel_Name']
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=d
    short=n.replace('void ','').replace('rocprim::ROCPRIM_400200_NS::detail::','rp::').replace('fb::(anonymous namespace)::','')
    if d>=float(sys.argv[1]) if len(sys.argv)>1 else True:
        print("%8.1f us  +%7.1f  %s" % (d, (int(r['Start_Timestamp'])-t0)/1e3, short[:90]))
    if 'k_tet_rest' in n: break
print("sum of kernel durations", tot)
