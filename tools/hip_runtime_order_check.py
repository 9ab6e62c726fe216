import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    print("torch sees gpu:", torch.cuda.is_available())
from euclider_amd import Parser
env = Parser().parse_file("scenes/3d_fresnel.json")
img = env.render((32, 32))
print("render ok", img.stats["rays"])
if mode == "lib_first":
    import torch
    print("torch sees gpu:", torch.cuda.is_available())
    x = torch.ones(4, device="cuda") * 2
    print(x.sum().item())
env.close()
