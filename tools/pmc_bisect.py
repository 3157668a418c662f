import sys, torch
st = sys.argv[1]
dev = torch.device('cuda', 0)
if st == 'A':
    x = torch.zeros(2_600_000_000, dtype=torch.bfloat16, device=dev); torch.cuda.synchronize(); print('A ok')
elif st == 'B':
    x = torch.zeros(1 << 28, dtype=torch.bfloat16).pin_memory(); y = x.to(dev, non_blocking=True); torch.cuda.synchronize(); print('B ok')
elif st == 'C':
    x = torch.zeros(1 << 20, dtype=torch.bfloat16, device=dev); torch.cuda.synchronize(); print('C ok')
elif st == 'D':
    sys.path.insert(0, '.')
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import mini_config
    u = AozoraUNet(mini_config(), dev); torch.cuda.synchronize(); print('D ok')
elif st == 'E':
    sys.path.insert(0, '.')
    import bench; x = torch.zeros(8, device=dev); torch.cuda.synchronize(); print('E ok')
elif st in ('F', 'G'):
    sys.path.insert(0, '.')
    import bench
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
    u = AozoraUNet(SDXL_BASE, dev); torch.cuda.synchronize(); print('F ok', flush=True)
    if st == 'G':
        bench.init_weights_on_device(u); torch.cuda.synchronize(); print('G ok')
elif st == 'H':
    sys.path.insert(0, '.')
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import mini_config
    from aozora_sdxl_training_amd.train_step import TrainStep
    pc = mini_config()
    u = AozoraUNet(pc, dev)
    step = TrainStep(u, mode='epsilon', grad_accum=1, use_graph=False)
    step.stream = torch.cuda.current_stream(); u.concurrent_wgrad = False
    B = 2
    l = step.micro_step(torch.randn(B, 4, 16, 16, device=dev).bfloat16(), torch.randn(B, 4, 16, 16, device=dev), torch.tensor([10, 500]),
                        torch.randn(B, 77, pc.cross_attention_dim, device=dev).bfloat16(), torch.randn(B, pc.pooled_dim, device=dev).bfloat16(),
                        torch.tensor([[128, 128, 0, 0, 128, 128]] * B, dtype=torch.bfloat16, device=dev))
    torch.cuda.synchronize(); print('H ok', l.item())
