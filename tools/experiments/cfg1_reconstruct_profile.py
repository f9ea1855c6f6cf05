"""cProfile of Scene.add_images in the configs[1] example (8 views of 512 x 384, synthetic network): where do the 9 s go?"""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import starst3r_amd as st
from st3r_synth.synth_model import SyntheticNetwork
net = SyntheticNetwork(n_views=8, width=512, height=384, seed=2)
imgs = net.images()
sc = st.Scene(device="cuda:0")
sc.add_images(net, imgs[:2])
pr = cProfile.Profile(); pr.enable()
sc.add_images(net, imgs[2:]); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
