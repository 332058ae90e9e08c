import faulthandler, gc, os, sys
faulthandler.enable()
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_model_gpu as T
mode = sys.argv[1]
T.test_phrase_prompt_through_graph_runtime()
print("phrase test done", flush=True)
if mode == "gc":
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache(); print("gc done", flush=True)
if mode == "sync":
    torch.cuda.synchronize()
T.test_software_pipelined_runtime(1)
print("pipelined test done", flush=True)
