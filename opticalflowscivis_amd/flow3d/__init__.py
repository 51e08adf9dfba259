"""Drop-in mirror of the reference's Flow-3D/ package layout (model/, train.py, inference_img.py)."""
