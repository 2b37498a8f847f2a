"""The option bag the reference's drivers define inline (train.ipynb cell 0, test.ipynb cell 0,
app.py:1-60).  Any object with these attributes works (`create_model` is duck-typed); this class only
saves callers — bench.py, the tests — from retyping the ~55 fields.  Values are the reference's
train.ipynb defaults except the paths."""


class Option(object):
    def __init__(self, **overrides):
        self.batchSize = 1
        self.fineSize = 256
        self.input_nc = 3
        self.input_nc_g = 6
        self.output_nc = 3
        self.ngf = 64
        self.ndf = 64
        self.which_model_netD = 'basic'
        self.which_model_netF = 'feature'
        self.which_model_netG = 'unet_ipsr'
        self.which_model_netP = 'unet_256'
        self.triple_weight = 1
        self.name = 'IPSR_inpainting'
        self.n_layers_D = '3'
        self.gpu_ids = [0]
        self.model = 'ipsr_net'
        self.checkpoints_dir = './checkpoints'
        self.norm = 'instance'
        self.fixed_mask = 1
        self.use_dropout = True
        self.init_type = 'normal'
        self.mask_type = 'random'
        self.lambda_A = 100
        self.threshold = 5 / 16.0
        self.stride = 1
        self.shift_sz = 1
        self.mask_thred = 1
        self.bottleneck = 512
        self.gp_lambda = 10.0
        self.ncritic = 5
        self.constrain = 'MSE'
        self.strength = 1
        self.init_gain = 0.02
        self.cosis = 1
        self.gan_type = 'lsgan'
        self.gan_weight = 0.2
        self.overlap = 4
        self.skip = 0
        self.display_freq = 1000
        self.print_freq = 50
        self.save_latest_freq = 5000
        self.save_epoch_freq = 1
        self.continue_train = False
        self.epoch_count = 1
        self.phase = 'train'
        self.which_epoch = ''
        self.niter = 20
        self.niter_decay = 100
        self.beta1 = 0.5
        self.lr = 0.0002
        self.lr_policy = 'lambda'
        self.lr_decay_iters = 50
        self.isTrain = True
        for k, v in overrides.items():
            setattr(self, k, v)
