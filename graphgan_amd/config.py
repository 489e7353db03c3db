"""Hyper-parameters of the GraphGAN trainer.

Same names, meaning and defaults as the reference's flag module (``src/GraphGAN/config.py:1-41``)
so that a user's existing ``config.py`` keeps working: ``graph_gan.py`` here imports a module
called ``config`` from the working directory first and falls back to this file.  Knobs that only
exist for the MI355X engine are prefixed ``engine_`` and default to reference behaviour.
"""
import os

modes = ["gen", "dis"]  # order of emb_filenames / results lines

# ---- training (reference config.py:4-16)
batch_size_gen = 64
batch_size_dis = 64
lambda_gen = 1e-5
lambda_dis = 1e-5
n_sample_gen = 20
lr_gen = 1e-3
lr_dis = 1e-3
n_epochs = 20
n_epochs_gen = 30
n_epochs_dis = 30
gen_interval = n_epochs_gen
dis_interval = n_epochs_dis
update_ratio = 1

# ---- model saving (reference config.py:19-20)
load_model = False
save_steps = 10

# ---- other hyper-parameters (reference config.py:23-25)
n_emb = 50
multi_processing = False  # kept for compatibility; the C++ tree builder is always threaded
window_size = 2

# ---- application / dataset / paths (reference config.py:28-41), relative to the working directory
app = "link_prediction"
dataset = "CA-GrQc"
_base = os.environ.get("GRAPHGAN_ROOT", "../..")
train_filename = _base + "/data/" + app + "/" + dataset + "_train.txt"
test_filename = _base + "/data/" + app + "/" + dataset + "_test.txt"
test_neg_filename = _base + "/data/" + app + "/" + dataset + "_test_neg.txt"
pretrain_emb_filename_d = _base + "/pre_train/" + app + "/" + dataset + "_pre_train.emb"
pretrain_emb_filename_g = _base + "/pre_train/" + app + "/" + dataset + "_pre_train.emb"
emb_filenames = [_base + "/results/" + app + "/" + dataset + "_gen_.emb",
                 _base + "/results/" + app + "/" + dataset + "_dis_.emb"]
result_filename = _base + "/results/" + app + "/" + dataset + ".txt"
cache_filename = _base + "/cache/" + dataset + ".pkl"
model_log = _base + "/log/"

# ---- engine-only knobs (no reference counterpart)
engine_seed = 0               # Philox key of the walk sampler + host RNG of root selection / batch shuffles
engine_optimizer = "adam_dense"  # "adam_dense" = TF1.8 semantics (parity); "adam_lazy" | "sgd" = scale modes
engine_device = 0
engine_tree_device = True    # BFS trees on the GPU (False: threaded host BFS, same trees)
engine_tree_threads = 0       # host BFS only; 0 = all host cores
engine_profile_every = 1      # HIP events on every k-th walk launch; 1 = every launch and pass (passes synchronous); 0 = none
engine_tree_budget_gb = 160.0  # all N BFS trees stay resident (reference :31-46) when they fit this much HBM; otherwise the epoch runs over root batches (gg_epoch_*)
engine_batch_roots = 0        # roots per batch of a root-batched epoch; 0 = what the budget holds (at most 16 384)
engine_emb_text = True        # write the reference's .emb text after every epoch (graph_gan.py:293-306); ~50 GB per write at N = 10^7, d = 256
engine_emb_sidecar = False    # also write <emb_filename>.bin: the same fp32 numbers in binary (utils.read_embeddings_bin)
