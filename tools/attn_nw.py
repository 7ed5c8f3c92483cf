import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'text-generation-inference_amd')
import microbench as mb
mb.bench_attn(32,32,32,128,1024)
mb.bench_attn(32,32,32,128,1024, ns=2)
