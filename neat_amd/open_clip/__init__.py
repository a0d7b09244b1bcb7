from .factory import (add_model_config, collect_params, create_model, create_model_and_transforms,
                      freeze_parameters, get_model_config, list_models, load_checkpoint)
from .model import CLIP, LayerNorm, VideoCLIP
from .tokenizer import SimpleTokenizer, get_tokenizer
from .transform import GpuImageTransform, image_transform
