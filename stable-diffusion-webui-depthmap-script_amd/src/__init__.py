"""MI355X-native hot path of stable-diffusion-webui-depthmap-script.

This directory mirrors the reference's ``src`` package for the modules on the hot path, so that code
written against the reference (``from src.stereoimage_generation import create_stereoimages``,
``from src.core import core_generation_funnel``) runs unchanged when this extension root is on sys.path.
"""
