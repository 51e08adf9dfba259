"""Drop-in mirror of the reference's Flow-2D/ package layout (model/, train.py, inference_img.py)."""
