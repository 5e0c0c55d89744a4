import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(131072,64,64),(131072,256,64),(131072,64,256),(131072,128,256),(32768,512,128),(32768,128,512),(32768,256,512),(8192,1024,256),(8192,256,1024),(8192,512,1024),(2048,2048,512),(2048,512,2048),(2048,1024,2048)]
for M,N,K in shapes:
    a = torch.randn(M,K,device='cuda'); w = torch.randn(N,K,device='cuda')
    for _ in range(3): y = a @ w.t()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1)/20*1e3
    print('M=%6d N=%5d K=%5d  %7.1f us  %6.1f TFLOP/s' % (M,N,K,t, 2.0*M*N*K/t/1e6))
