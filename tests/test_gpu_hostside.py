"""Host-side parity tests that need no GPU, re-collected under the `gpu` marker so that the driver's GPU-box run
(`pytest -m gpu`) executes them as well: the C++ trie tokenizer, clip_ar_xform and the checkpoint mapping against the
reference's vectors, and the world_size-2 gloo tests of the collective glue.  (pytest collects imported test functions;
the module-level mark applies to all of them.)"""
import pytest

pytestmark = pytest.mark.gpu

from tests.test_host_cpu import (  # noqa: E402,F401
    test_trie_tokenizer_matches_reference_golden, test_stack_batch_matches_reference_golden,
    test_tensorize_tail_matches_reference_xform, test_cabi_exports_every_declared_symbol,
    test_optimizer_arguments_reach_the_optimizer)
from tests.test_checkpoint_xform_cpu import (  # noqa: E402,F401
    test_clip_ar_xform_matches_reference, test_reference_checkpoint_document_layout,
    test_optimizer_state_maps_into_flat_buffers_and_back)
from tests.test_distributed_gloo import (  # noqa: E402,F401
    test_all_gather_matches_reference_vectors, test_row_sharded_infonce_equals_global_loss,
    test_grad_buckets_cover_the_flat_buffer)
