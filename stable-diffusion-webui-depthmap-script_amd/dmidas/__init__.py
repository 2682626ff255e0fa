"""Drop-in for the hot path of the reference's ``dmidas`` package (MiDaS 3.1 DPT): ``dmidas.dpt_depth.DPTDepthModel``
with the BEiT backbones, MI355X-first (src/vit_mi355x.py).  Checkpoint key names are the reference's."""
