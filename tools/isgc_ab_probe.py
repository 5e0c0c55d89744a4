import sys, copy, torch
sys.path.insert(0, '/root/repo')
import humaniflow_amd as H
from humaniflow_amd.train import train_step, synthetic_train_batch, GradBucketReducer
from humaniflow_amd.losses import HumaniflowLoss
from humaniflow_amd.synthetic import init_benchmark_weights
dev = torch.device('cuda:0')
cfg = H.get_humaniflow_cfg_defaults(); cfg.MODEL.NUM_RESNET_LAYERS = 18
def run(isgc):
    torch.manual_seed(0)
    model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).to(dev); model.train()
    model.isgc_kernel = isgc
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=72, create_transl=False).to(dev)
    for p in smpl.parameters(): p.requires_grad_(False)
    crit = HumaniflowLoss(cfg.LOSS, cfg.DATA.PROXY_REP_SIZE)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=cfg.TRAIN.LR, fused=True)
    red = GradBucketReducer(params, None)
    batches = [synthetic_train_batch(smpl, 72, cfg.DATA.PROXY_REP_SIZE, dev, seed=1000 + i, data_cfg=cfg.DATA) for i in range(3)]
    torch.manual_seed(1234)
    losses = []
    snap = None
    for k in range(8):
        loss, sk = train_step(model, smpl, crit, opt, red, *batches[k % 3], 8)
        losses.append(loss.item())
        if k == 0:
            snap = {n: p.detach().clone() for n, p in model.named_parameters()}
    return losses, snap
l1, s1 = run(True); l0, s0 = run(False)
print('isgc=1', ['%.3f' % v for v in l1]); print('isgc=0', ['%.3f' % v for v in l0])
worst = sorted(((s1[n] - s0[n]).abs().max().item(), n) for n in s1)[-6:]
print('largest parameter differences after step 1 (lr = %g):' % cfg.TRAIN.LR); [print('  %.3e %s' % w) for w in worst]
