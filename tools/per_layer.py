import csv,glob,os,sys
f=max(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'), key=os.path.getmtime)   # gpurun_out/ accumulates runs: the newest one
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_nchw_to_nhwc' in r['Kernel_Name']]
step=rows[idx[-2]:idx[-1]]
enc=[(8,64,64,256),(64,128,128,128),(128,256,256,64),(256,512,512,32),(512,512,512,16)]
convs=[]
for cin,cmid,cout,hw in enc: convs+= [(cin,cmid,hw),(cmid,cout,hw)]
dec=[(1024,512,256,32),(512,256,128,64),(256,128,64,128),(128,64,64,256)]
for cin,cmid,cout,hw in dec: convs+=[(cin,cmid,hw),(cmid,cout,hw)]
B=16
fw=[r for r in step if 'k_conv3x3_' in r['Kernel_Name'] or 'k_wgrad_bf16' in r['Kernel_Name'] or 'k_wgrad_f32' in r['Kernel_Name']]
def dur(r): return (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
tf=tw=td=0
for i,r in enumerate(fw[:18]):
    cin,cout,hw=convs[i]; fl=2*9*cin*cout*hw*hw*B; d=dur(r); tf+=d
    print(f"fwd  {i:2d} {cin:5d}->{cout:4d} @{hw:3d} {r['Kernel_Name'][14:36]:22s} {d:7.1f} us {fl/d/1e6:7.1f} TF  wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
# backward: the weight-gradient chain runs on a side stream, so start-time order interleaves the two kernels of a conv
# arbitrarily; classify by kernel name (each stream keeps its own launch order: conv 17 first)
bw_w=[r for r in fw[18:] if 'k_wgrad' in r['Kernel_Name']]
bw_d=[r for r in fw[18:] if 'k_wgrad' not in r['Kernel_Name']]
assert len(bw_w)==18 and len(bw_d)==17, (len(bw_w), len(bw_d))
kw=kd=0
for i in reversed(range(18)):
    cin,cout,hw=convs[i]; fl=2*9*cin*cout*hw*hw*B
    r=bw_w[kw]; kw+=1; d=dur(r); tw+=d
    s=f"bwd  {i:2d} {cin:5d}->{cout:4d} @{hw:3d} wgrad {r['Kernel_Name'][:40].split('(')[0][-20:]:20s} {d:7.1f} us {fl/d/1e6:7.1f} TF wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}"
    if i!=0:
        r=bw_d[kd]; kd+=1; d=dur(r); td+=d
        s+=f" | dgrad {r['Kernel_Name'][14:36]:22s} {d:7.1f} us {fl/d/1e6:7.1f} TF wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}"
    print(s)
print("fwd us",tf,"wgrad us",tw,"dgrad us",td)
tot=(int(step[-1]['End_Timestamp'])-int(step[0]['Start_Timestamp']))/1e3
busy=sum(dur(r) for r in step)
print("step span us",tot,"busy us",busy,"launches",len(step))
