from .trace_mistral import TraceMistralForCausalLM  # noqa: F401
from ..config import TraceConfig as TraceMistralConfig  # noqa: F401
