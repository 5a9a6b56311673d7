import sys
sys.path.insert(0,'scratch')
from dbtrace import rows
SH=[("g", 18432, 320, 320), ("g", 18432, 2560, 320), ("g", 18432, 960, 320), ("g", 18432, 320, 1280),
          ("g", 4608, 640, 640), ("g", 4608, 5120, 640), ("g", 4608, 1920, 640), ("g", 1152, 10240, 1280), ("g", 1152, 3840, 1280),
          ("c", 2, 96, 320, 320), ("c", 2, 96, 640, 320), ("c", 2, 48, 640, 640), ("c", 2, 48, 1280, 640), ("c", 2, 24, 1280, 1280),
          ("c", 1, 384, 256, 256), ("c", 1, 192, 512, 512), ("c", 1, 768, 128, 128)]
def fl(s):
    return 2.0*s[1]*s[2]*s[3] if s[0]=="g" else 2.0*s[1]*s[2]*s[2]*s[3]*9*s[4]
res={}
for path in sys.argv[1:]:
    r=rows(path)
    # group: each shape = 3 x (igemm [+ reduce]); walk sequentially
    ig=[]; cur=None
    for x in r:
        if 'igemm_kernel' in x[0] or 'conv_win_kernel' in x[0]:
            cur=[x[0],x[1],x[3],x[4]]; ig.append(cur)
        elif 'splitk_reduce' in x[0] and cur is not None:
            cur[3]+=x[4]
    assert len(ig)==3*len(SH),(len(ig),path)
    res[path]=[min(v[3] for v in ig[3*i:3*i+3]) for i in range(len(SH))]
    res[path+"n"]=[ig[3*i][0][13:40]+" z%d"%ig[3*i][2] for i in range(len(SH))]
for i,s in enumerate(SH):
    print("%-28s"%str(s), "  ".join("%7.1fus %4.0fTF"%(res[p][i], fl(s)/res[p][i]/1e6) for p in sys.argv[1:]))
