// Stateful decode session (KV-cached fast path).
#pragma once
#include "engine.h"

struct wb_session;
