class VGG16_Weights:
    DEFAULT = None


def vgg16(*args, **kwargs):
    raise RuntimeError("torchvision.models.vgg16 is stubbed in the oracle harness")
