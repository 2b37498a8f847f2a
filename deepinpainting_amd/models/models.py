"""create_model — entry of the drop-in boundary (reference models/models.py:2-12)."""


def create_model(opt):
    model = None
    print(opt.model)
    if opt.model == 'ipsr_net':
        from .IPSR import IPSR
        model = IPSR()
    else:
        raise ValueError("Model [%s] not recognized." % opt.model)
    model.initialize(opt)
    print("model [%s] was created" % (model.name()))
    return model
